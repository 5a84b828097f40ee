cd $GRAFT_REPO_ROOT; O=gpurun_out/s12; mkdir -p $O
for single in 127 126 125 123 119 111 95 63 0; do
  echo -n "REPMODE_DET_SINGLE=$single: "
  REPMODE_DET_SINGLE=$single timeout 300 python -m pytest tests/test_bf16_end_to_end_gpu.py -m gpu -q -k "deterministic" 2>&1 | tail -1
done | tee $O/det_sites.log
