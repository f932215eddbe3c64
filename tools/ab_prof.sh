cd /tmp; export TMPDIR=/tmp
for L in libmm_prev.so libmm_manifolds.so; do
  export MM_MANIFOLDS_LIB=/root/repo/matrix-manifolds_amd/lib/$L
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ab_$L -o s -- python3 /root/repo/bench.py --no-cpu-baseline --steps 30 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/ab_$L.log 2>&1
  echo == $L; tail -1 $GRAFT_REPO_ROOT/gpurun_out/ab_$L.log | cut -c1-200
  cut -d, -f1-4 $GRAFT_REPO_ROOT/gpurun_out/ab_$L/s_kernel_stats.csv | cut -c1-50,140-300 | head -8
done
