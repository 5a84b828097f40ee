# HBM traffic per kernel family from rocprofv3 PMC passes (one counter per pass), run through gpurun:
#   bash tools/pmc_traffic.sh ; python profiles/pmc_summary.py gpurun_out > profiles/rNN_pmc_traffic.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $R/gpurun_out/pmc_$c.log 2>&1
  tail -1 $R/gpurun_out/pmc_$c.log | cut -c1-120
done
