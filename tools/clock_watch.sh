# Samples the GPU clocks / power while a bench runs (is the pipelined step clock- or power-limited?)
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/clock_watch_$1.txt 2>&1 &
W=$!
shift
"$@"
kill $W 2>/dev/null
