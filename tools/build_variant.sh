# usage: tools/build_variant.sh <name> "<extra hipcc flags>" [object ...]   (run in the build container)
# Builds ab_libs/libexcenv_<name>.so: the listed translation units (default: env_pmsm) recompiled with the extra flags, every
# other object taken from the in-tree build. For same-session A/B runs on the GPU box (EXCENV_HIP_LIB=...).
set -e
NAME=$1; FLAGS=$2; shift 2
OBJS=${@:-env_pmsm}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/exciting-environments_amd/csrc
TMP=/tmp/variant_$NAME; mkdir -p $TMP $ROOT/ab_libs
ALL="excenv_api transpose calib env_pendulum env_msd env_cartpole env_acrobot env_tank env_pmsm env_pmsm_sat"
LINK=""
for o in $ALL; do
  if echo " $OBJS " | grep -q " $o "; then
    EXTRA="-fno-slp-vectorize"
    (cd $SRC && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $EXTRA $FLAGS -c $o.hip -o $TMP/$o.o) &
    LINK="$LINK $TMP/$o.o"
  else
    LINK="$LINK $SRC/$o.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab_libs/libexcenv_$NAME.so $LINK
echo built $ROOT/ab_libs/libexcenv_$NAME.so
