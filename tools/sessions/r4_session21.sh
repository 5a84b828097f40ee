# helper kernels of the per-expert levels (box sums in one phase sequence, expert_mix grid / index arithmetic): tests, whole step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s21; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "box or expert_mix or unmerged or per_expert or mix" 2>&1 | tail -2
BASE_VARIANT=k2old bash tools/sessions/r4_session20.sh
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/bench.py --no-cpu-baseline --no-prof --no-fwd --steps 10 --warmup 3 > $O/trace.log 2>&1
python $R/tools/trace_by_grid.py $O/tr 'box_|expert_mix|gemm3|k2s2_wgrad' 30 | cut -c1-50,70-
rm -rf $O/tr
