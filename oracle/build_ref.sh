#!/bin/sh
# Compile the reference's own `default` CPU kernels, from the sources where they
# lie under /root/reference, into oracle/_ref/librefkernels.so (double) and
# librefkernels_sp.so (single, -DRTE_USE_SP: rte/kernels/mo_rte_kind.F90:32-36).
# This does NOT run the reference's CMake; it is the nine kernel files compiled
# directly with AMD flang (the only Fortran compiler in the image), plus our own
# oracle/ref_wrappers.F90 (C-callable wrappers of the two routines without a C binding).
# Outputs are binaries only and are git-ignored.  Test infrastructure, not product code.
# A failed compile of any file stops the build: a partial library is never linked.
set -e
R=${REFERENCE_ROOT:-/root/reference}
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
SRCS="rte/kernels/mo_rte_kind.F90 rte/kernels/mo_rte_util_array.F90 rte/kernels/mo_rte_solver_kernels.F90 \
rte/kernels/mo_fluxes_broadband_kernels.F90 rte/kernels/mo_optical_props_kernels.F90 \
rte/kernels/mo_gas_optics_constants.F90 rte/kernels/mo_gas_optics_utils.F90 \
rrtmgp/kernels/mo_gas_optics_rrtmgp_kernels.F90 rrtmgp/kernels/mo_cloud_optics_rrtmgp_kernels.F90"
for prec in dp sp; do
  B=$OUT/build_$prec
  rm -rf "$B"; mkdir -p "$B"
  DEF=""; SUF=""
  if [ $prec = sp ]; then DEF="-DRTE_USE_SP"; SUF="_sp"; fi
  (
    cd "$B"
    for f in $SRCS; do
      $FC -O2 -fPIC $DEF -c "$R/$f" 2> "$B/err.log" || { echo "build_ref: $f failed:" >&2; cat "$B/err.log" >&2; exit 1; }
    done
    $FC -O2 -fPIC $DEF -c "$HERE/ref_wrappers.F90" 2> "$B/err.log" || { cat "$B/err.log" >&2; exit 1; }
    $FC -shared -o "$OUT/librefkernels$SUF.so" ./*.o
  )
  rm -rf "$B"
done
ls -l "$OUT"
