cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
python tools/attn_ab.py AB_SHAPE=16x64x2048 AB_SHAPE=16x64x2048,AB_NORM=1 AB_SHAPE=16x64x2048,AB_FLAGS=3 AB_SHAPE=4x24x8704,AB_FLAGS=3 AB_SHAPE=16x24x2560,AB_FLAGS=3 2>&1 | tee $O/attn_guide_shape.txt
python tools/attn_ab.py --fp8 AB_SHAPE=16x64x2048 AB_SHAPE=4x24x8704 2>&1 | tee $O/attn_guide_shape_fp8.txt
