#!/bin/bash
# Phase times inside the loader-wave tiles: builds a DEBUG copy of the library (conv_x3.hip with -DIPRGAN_X3WS_TIMING: s_memtime stamps
# of multiplying wave 0 of every block into a device array) and runs scripts/probe/ws_phase_times.py on it.  The stamps cost ~20 % of
# the launch time: read the numbers as shares, not as absolute times.  X3WS_EXTRA=-DIPRGAN_X3WS_HALFDMA: the loaders skip every other
# piece (results are garbage) - tells whether the K step is bound by the CU's LDS-DMA intake.  Here: bash scripts/probe/ws_phase_times.sh build ; gpurun: ... run
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
if [ "$1" = build ]; then
  T=$(mktemp -d); cd $R/ipr-gan_amd/csrc; make -j6 > /dev/null
  for f in conv_igemm conv_pipe wgrad_halo wgrad_x3 norm spectral elementwise ssim comm; do cp $f.o $T/; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -DIPRGAN_X3WS_TIMING $X3WS_EXTRA -c conv_x3.hip -o $T/conv_x3.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $T/*.o -ldl -o $R/ipr-gan_amd/iprgan/libiprgan_dbg.so; rm -rf $T
else
  export IPRGAN_LIB=$R/ipr-gan_amd/iprgan/libiprgan_dbg.so
  for T in 36 37 35; do timeout 300 python $R/scripts/probe/ws_phase_times.py $T 2>&1 | grep '^{'; done
fi
