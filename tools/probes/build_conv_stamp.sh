#!/bin/bash
# Diagnostic build of the library with s_memtime stamps in conv_halo_kernel (-DSF_STAMP): tools/probes/abl/libabl_cstamp.so,
# used by tools/probes/conv_stamp.py.  The shipped library is built without the macro and executes no stamp.
set -e
cd "$(dirname "$0")/../../self-forcing_amd/csrc"
make -s
mkdir -p ../../tools/probes/abl
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -DSF_STAMP $SF_EXTRA -c conv_halo.hip -o /tmp/abl_conv_stamp.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/abl_conv_stamp.o gemm_bf16.o attention.o elementwise.o \
   small_linear.o dit_forward.o conv_igemm.o vae_elementwise.o vae_decode.o t5_encoder.o capi.o -o ../../tools/probes/abl/libabl_cstamp${SF_TAG}.so
echo built conv stamp
