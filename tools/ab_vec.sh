cd /tmp; export TMPDIR=/tmp
for L in "$@"; do
  export MM_MANIFOLDS_LIB=/root/repo/matrix-manifolds_amd/lib/$L
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/vab_$L -o s -- python3 /root/repo/tools/bench_configs.py --only lorentz11_f32_gram > /dev/null 2>&1
  echo == $L; grep "gram_bwd" $GRAFT_REPO_ROOT/gpurun_out/vab_$L/s_kernel_stats.csv | awk -F, '{print $(NF-6), $(NF-5), $(NF-4)}'
done
