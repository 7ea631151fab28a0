"""Drop-in alias for the inference half of the reference's src/train/model.py -> loongx_amd.train.model (MI355X)."""
from loongx_amd.train.model import *  # noqa: F401,F403
from loongx_amd.train.model import OminiModel, DUAN, FeaturePyramidPooling, EEGEncoder, PPGEncoder, FNIRSEncoder, MotionEncoder  # noqa: F401
