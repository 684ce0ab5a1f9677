#!/bin/sh
# Build oracle/_ref/bin/ref_load_driver: the reference's own frontend (default CPU kernels, oracle/_ref/librefkernels.so)
# plus oracle/ref_load_driver.F90, so that the reference's ty_gas_optics_rrtmgp%load can be run on a raw table and its
# results compared with rte-rrtmgp_amd/kdist_load.py (tests/test_kdist_load.py).  Needs /root/reference + flang; outputs are
# binaries only under oracle/_ref/ (git-ignored).  Test infrastructure.
set -e
R=${REFERENCE_ROOT:-/root/reference}
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
OUT=$HERE/_ref
B=$OUT/load_build
rm -rf "$B"; mkdir -p "$B" "$OUT/bin"
FFLAGS="-O1 -fPIC -DAMDFLANG_WORKAROUND"
SRCS="rte/kernels/mo_rte_kind.F90 rte/kernels/api/mo_rte_util_array.F90 rte/kernels/mo_gas_optics_constants.F90 \
rte/kernels/api/mo_fluxes_broadband_kernels.F90 rte/kernels/api/mo_gas_optics_utils.F90 \
rte/kernels/api/mo_optical_props_kernels.F90 rte/kernels/api/mo_rte_solver_kernels.F90 \
rrtmgp/kernels/api/mo_gas_optics_rrtmgp_kernels.F90 rrtmgp/kernels/api/mo_cloud_optics_rrtmgp_kernels.F90 \
rte/frontend/mo_rte_config.F90 rte/frontend/mo_rte_util_array_validation.F90 rte/frontend/mo_optical_props.F90 \
rte/frontend/mo_source_functions.F90 rte/frontend/mo_fluxes.F90 \
rte/frontend/gas-optics-template/mo_gas_optics_util_string.F90 rte/frontend/gas-optics-template/mo_gas_concentrations.F90 \
rte/frontend/gas-optics-template/mo_gas_optics.F90 rrtmgp/frontend/mo_gas_optics_rrtmgp.F90"
cd "$B"
OBJS=""
for f in $SRCS; do
  $FC $FFLAGS -c "$R/$f" 2> err.log || { echo "build_load_check: $f failed:" >&2; cat err.log >&2; exit 1; }
  OBJS="$OBJS $(basename "$f" .F90).o"
done
$FC $FFLAGS -c "$ROOT/shim/rte_hip_fortran_shim.F90" -o shim.o 2> err.log || { cat err.log >&2; exit 1; }
$FC $FFLAGS -c "$HERE/mo_raw_stream.F90" 2> err.log || { cat err.log >&2; exit 1; }
$FC $FFLAGS -c "$HERE/ref_load_driver.F90" 2> err.log || { cat err.log >&2; exit 1; }
gcc -O1 -fPIC -c "$HERE/abi_recorder.c" -o abi_recorder.o
$FC -o "$OUT/bin/ref_load_driver" ref_load_driver.o mo_raw_stream.o $OBJS shim.o abi_recorder.o -L"$OUT" -lrefkernels -L"$HERE" -loracle \
    -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,'$ORIGIN/../..'
cd "$OUT"; rm -rf "$B"
ls -l "$OUT/bin/ref_load_driver"
