# same-box A/B of the dominant kernel between the product library and a variant build, e.g.
#   REPMODE_EXTRA_FLAGS=-DRM_CONV_SCHED REPMODE_OUT=$PWD/repmode_amd/librepmode_hip_sched.so \
#     REPMODE_BUILD_DIR=$PWD/repmode_amd/csrc/build_sched bash repmode_amd/csrc/build.sh      (here, cross-compiled)
#   gpurun -- 'bash tools/ab_variant.sh repmode_amd/librepmode_hip_sched.so'
V=$GRAFT_REPO_ROOT/$1
for rep in 1 2; do
  for lib in "" $V; do
    echo "lib=${lib:-product}"
    REPMODE_LIB=$lib timeout 60 python $GRAFT_REPO_ROOT/tools/conv_microbench.py 32 32 32 64 64 1000 2>&1 | tail -1
  done
done
REPMODE_LIB= timeout 60 python $GRAFT_REPO_ROOT/tools/conv_microbench.py 64 32 32 64 64 600 2>&1 | tail -1
REPMODE_LIB=$V timeout 60 python $GRAFT_REPO_ROOT/tools/conv_microbench.py 64 32 32 64 64 600 2>&1 | tail -1
