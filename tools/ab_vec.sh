cd /tmp; export TMPDIR=/tmp
for T in 1 2 4 8 16; do
  export MM_GRAM_BWD_TPW=$T
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/vab_$T -o s -- python3 /root/repo/tools/bench_configs.py --only lorentz11_f32_gram > /dev/null 2>&1
  echo == tpw $T; grep "gram_bwd" $GRAFT_REPO_ROOT/gpurun_out/vab_$T/s_kernel_stats.csv | awk -F, '{print $(NF-6), $(NF-5), $(NF-4)}'
done
