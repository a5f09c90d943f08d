cd /root/repo
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
rm -rf gpurun_out/pprof; mkdir -p gpurun_out/pprof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pprof/r -o kt -- python3 bench.py --cpu-sample 0 --single-frames 0 --steps 200 --warmup 20 > gpurun_out/pprof/r.log 2>&1
f=$(find gpurun_out/pprof/r -name "*kernel_trace.csv" | head -1)
python3 tools/trace_overlap.py $f
tail -1 gpurun_out/pprof/r.log | cut -c1-200
