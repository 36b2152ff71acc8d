# kernel timeline of one training step (rocprofv3 --kernel-trace): bash tests/tools/run_timeline.sh [bf16|f32] [extra bench flags]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/tl
which=${1:-bf16}; shift
flags=""; [ "$which" = "bf16" ] && flags="--precision bf16"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/$which -o t -- python3 bench.py $flags "$@" --no_cpu_baseline --no_native_leg --steps 6 --warmup 4 --prof_kind 0 > gpurun_out/tl/bench_$which.json 2> gpurun_out/tl/err.txt
f=$(find gpurun_out/tl/$which -name "*kernel_trace.csv" | head -1)
python tests/tools/timeline.py $f 2 > gpurun_out/tl/timeline_$which.txt
find gpurun_out/tl -name "*.csv" -delete; find gpurun_out/tl -name "*.db" -delete
tail -4 gpurun_out/tl/timeline_$which.txt; cut -c1-200 gpurun_out/tl/bench_$which.json
