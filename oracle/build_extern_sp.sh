#!/bin/sh
# The single-precision counterpart of oracle/build_extern.sh (the reference's RTE_ENABLE_SP / -DRTE_USE_SP build,
# rte/kernels/mo_rte_kind.F90:32-36): the reference's UNCHANGED frontend compiled with wp = single, linked against
# librte_rrtmgp_hip_sp.so (+ the shim) and against the reference's own single-precision CPU kernels
# (oracle/_ref/librefkernels_sp.so).  Programs: the reference's three data-free unit tests and oracle/ref_frontend_driver.F90,
# as oracle/_ref/bin/<name>_sp and <name>_sp_cpuref.  Outputs are binaries only, under oracle/_ref/ (git-ignored).
set -e
R=${REFERENCE_ROOT:-/root/reference}
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
LIBDIR=$ROOT/rte-rrtmgp_amd
OUT=$HERE/_ref
B=$OUT/extern_build_sp
rm -rf "$B"; mkdir -p "$B" "$OUT/bin"
FFLAGS="-O2 -fPIC -DAMDFLANG_WORKAROUND -DRTE_USE_SP"
API="rte/kernels/mo_rte_kind.F90 rte/kernels/api/mo_rte_util_array.F90 rte/kernels/mo_gas_optics_constants.F90 \
rte/kernels/api/mo_fluxes_broadband_kernels.F90 rte/kernels/api/mo_gas_optics_utils.F90 \
rte/kernels/api/mo_optical_props_kernels.F90 rte/kernels/api/mo_rte_solver_kernels.F90 \
rrtmgp/kernels/api/mo_gas_optics_rrtmgp_kernels.F90 rrtmgp/kernels/api/mo_cloud_optics_rrtmgp_kernels.F90"
FRONT="rte/frontend/mo_rte_config.F90 rte/frontend/mo_rte_util_array_validation.F90 rte/frontend/mo_optical_props.F90 \
rte/frontend/mo_source_functions.F90 rte/frontend/mo_fluxes.F90 rte/frontend/mo_rte_lw.F90 rte/frontend/mo_rte_sw.F90 \
rte/frontend/gas-optics-template/mo_gas_optics_util_string.F90 rte/frontend/gas-optics-template/mo_gas_concentrations.F90 \
rte/frontend/gas-optics-template/mo_gas_optics.F90 \
rrtmgp/frontend/mo_gas_optics_rrtmgp.F90 rrtmgp/frontend/mo_cloud_optics_rrtmgp.F90 \
rrtmgp/frontend/mo_aerosol_optics_rrtmgp_merra.F90 rte/extensions/mo_fluxes_byband.F90"
TESTUTIL="examples/shared-utils/mo_testing_utils.F90 tests/mo_comparisons.F90"
cd "$B"
for f in $API $FRONT $TESTUTIL; do
  $FC $FFLAGS -c "$R/$f" 2> err.log || { echo "build_extern_sp: $f failed:" >&2; cat err.log >&2; exit 1; }
done
$FC $FFLAGS -c "$ROOT/shim/rte_hip_fortran_shim.F90" -o shim.o 2> err.log || { cat err.log >&2; exit 1; }
FRONT_OBJS=$(for f in $API $FRONT; do echo "$(basename "$f" .F90).o"; done)
HIPLINK="-L$LIBDIR -lrte_rrtmgp_hip_sp -Wl,-rpath,\$ORIGIN/../../../rte-rrtmgp_amd -Wl,-rpath,/opt/rocm/lib"
CPULINK="-L$OUT -lrefkernels_sp -L$HERE -loracle_sp -Wl,-rpath,\$ORIGIN/.. -Wl,-rpath,\$ORIGIN/../.."
for t in rte_lw_solver_unit_tests rte_sw_solver_unit_tests rte_optic_prop_unit_tests; do
  $FC $FFLAGS -c "$R/tests/$t.F90" 2> err.log || { echo "build_extern_sp: $t failed:" >&2; cat err.log >&2; exit 1; }
  $FC -o "$OUT/bin/${t}_sp" $t.o mo_comparisons.o mo_testing_utils.o $FRONT_OBJS shim.o $HIPLINK
  [ -f "$OUT/librefkernels_sp.so" ] && $FC -o "$OUT/bin/${t}_sp_cpuref" $t.o mo_comparisons.o mo_testing_utils.o $FRONT_OBJS shim.o $CPULINK
done
$FC $FFLAGS -c "$HERE/mo_raw_stream.F90" 2> err.log || { cat err.log >&2; exit 1; }
$FC $FFLAGS -c "$HERE/ref_frontend_driver.F90" 2> err.log || { cat err.log >&2; exit 1; }
$FC -o "$OUT/bin/ref_frontend_driver_sp" ref_frontend_driver.o mo_raw_stream.o $FRONT_OBJS shim.o $HIPLINK
[ -f "$OUT/librefkernels_sp.so" ] && $FC -o "$OUT/bin/ref_frontend_driver_sp_cpuref" ref_frontend_driver.o mo_raw_stream.o $FRONT_OBJS shim.o $CPULINK
cd "$OUT"; rm -rf "$B"
ls "$OUT/bin" | grep _sp
