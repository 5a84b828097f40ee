# A variant build of both libraries next to the product ones (for same-box A/Bs):
#   bash tools/build_variant.sh <name> "<extra hipcc flags>" [source.hip ...]
# -> variants/<name>/librepmode_hip.so + librepmode_torch.so; objects of the sources NOT named are reused from the product
# build (name the files the flags touch; none named = rebuild everything).  Run with
#   REPMODE_LIB=$GRAFT_REPO_ROOT/variants/<name>/librepmode_hip.so REPMODE_TORCH_LIB=$GRAFT_REPO_ROOT/variants/<name>/librepmode_torch.so
set -e
cd "$(dirname "$0")/.."; N=$1; F=$2; shift 2
D=variants/$N; mkdir -p $D/build
if [ $# -gt 0 ]; then
  cp -p repmode_amd/csrc/build/*.o $D/build/
  for s in "$@"; do rm -f $D/build/$(basename ${s%.hip}).o; done
fi
REPMODE_EXTRA_FLAGS="$F" REPMODE_OUT=$PWD/$D/librepmode_hip.so REPMODE_BUILD_DIR=$PWD/$D/build bash repmode_amd/csrc/build.sh
rm -rf $D/build
