#!/bin/bash
# tools/exp_build.sh N... -- build libpirip_hip.so variants with -DPIRIP_EXP=N (timing experiments on the headline instance only)
# into pirip_amd/lib_exp/N/.  The other objects come from the normal build.
set -e
cd "$(dirname "$0")/../pirip_amd/csrc"
make -s -j8 ../lib/libpirip_hip.so
for n in "$@"; do
  mkdir -p ../lib_exp/$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -DPIRIP_EXP=$n \
     -DPIRIP_WAVE_PROBE=3 -DPIRIP_WAVE_PROBE_P=24 $EXPFLAGS -c fsk_demod_wave.hip -o ../lib_exp/$n/fsk_demod_wave.o &
done
wait
for n in "$@"; do
  objs=$(ls ../lib/obj/*.o | grep -v fsk_demod_wave.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_exp/$n/libpirip_hip.so $objs ../lib_exp/$n/fsk_demod_wave.o
done
ls -la ../lib_exp/*/libpirip_hip.so | wc -l
