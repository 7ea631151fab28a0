# usage: bash tools/power_sample.sh <outfile> -- <command...> : samples rocm-smi power / clocks every 0.2 s while the command runs
OUT=$1; shift; shift
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge|memory)" | tr '\n' '|' ; echo; sleep 0.2; done ) > $OUT &
SP=$!
"$@"
kill $SP
