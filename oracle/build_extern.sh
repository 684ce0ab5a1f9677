#!/bin/sh
# Extern-mode link check of the reference's UNCHANGED Fortran frontend against librte_rrtmgp_hip.so, and the
# reference's own data-free unit-test programs built that way.
#
# What the reference's CMake does for -DRTE_KERNEL_MODE=extern (rte/kernels/CMakeLists.txt:3-13,
# rrtmgp/kernels/CMakeLists.txt:3-9) is done here by hand with flang: the kernel INTERFACE modules
# (rte/kernels/api/*.F90, rrtmgp/kernels/api/*.F90), then the frontend (rte/frontend, gas-optics-template,
# rrtmgp/frontend), all compiled from the sources where they lie under /root/reference; then
#   1. every undefined rte_* / rrtmgp_* / zero_array_* / ... symbol of those objects must be exported by
#      librte_rrtmgp_hip.so, the two Fortran externals (get_layer_mass_, get_layer_number_) by shim/rte_hip_fortran_shim.F90;
#   2. tests/rte_lw_solver_unit_tests.F90, rte_sw_solver_unit_tests.F90, rte_optic_prop_unit_tests.F90 (they need only
#      mo_testing_utils and mo_comparisons -- no netCDF, no data files) are linked against the HIP library and the shim
#      into oracle/_ref/bin/.  Running them (on a GPU box: tests/test_extern_frontend.py) drives the real rte_lw /
#      rte_sw / optical-props classes, with HOST arrays, through the library's staging path;
#   3. oracle/ref_frontend_driver.F90 (our driver of the reference's load -> gas_optics -> rte_lw / rte_sw) is linked
#      once against the HIP library and once against the reference's CPU kernels.
# Outputs are binaries only, under oracle/_ref/ (git-ignored, travels to the GPU box).  Nothing is copied.
set -e
R=${REFERENCE_ROOT:-/root/reference}
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
LIBDIR=$ROOT/rte-rrtmgp_amd
OUT=$HERE/_ref
B=$OUT/extern_build
rm -rf "$B"; mkdir -p "$B" "$OUT/bin"
# -DAMDFLANG_WORKAROUND: what the reference's CMake adds for flang (CMakeLists.txt:60-64)
FFLAGS="-O2 -fPIC -DAMDFLANG_WORKAROUND"
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
  $FC $FFLAGS -c "$R/$f" 2> err.log || { echo "build_extern: $f failed:" >&2; cat err.log >&2; exit 1; }
done
$FC $FFLAGS -c "$ROOT/shim/rte_hip_fortran_shim.F90" -o shim.o 2> err.log || { cat err.log >&2; exit 1; }
# ---- 1. symbol check
FRONT_OBJS=$(for f in $API $FRONT; do echo "$(basename "$f" .F90).o"; done)
nm -u $FRONT_OBJS | awk '$1=="U"{print $2}' | sort -u > undefined.txt
nm -D --defined-only "$LIBDIR/librte_rrtmgp_hip.so" | awk '{print $3}' | sort -u > exported.txt
nm --defined-only $FRONT_OBJS shim.o | awk 'NF==3{print $3}' | sort -u > defined_here.txt
MISSING=$(comm -23 undefined.txt exported.txt | comm -23 - defined_here.txt | grep -E '^(rte_|rrtmgp_|zero_array|set_to_scalar|net_byband|get_layer)' || true)
KERN=$(comm -12 undefined.txt exported.txt | grep -cE '^(rte_|rrtmgp_|zero_array|set_to_scalar|net_byband)')
echo "extern link check: $KERN kernel symbols referenced by the frontend are exported by librte_rrtmgp_hip.so"
if [ -n "$MISSING" ]; then echo "extern link check: MISSING symbols:" >&2; echo "$MISSING" >&2; exit 1; fi
echo "$KERN" > "$OUT/extern_symbols_ok.txt"
comm -12 undefined.txt exported.txt | grep -E '^(rte_|rrtmgp_|zero_array|set_to_scalar|net_byband)' >> "$OUT/extern_symbols_ok.txt"
# ---- 2. the reference's data-free unit-test programs, extern mode
for t in rte_lw_solver_unit_tests rte_sw_solver_unit_tests rte_optic_prop_unit_tests; do
  $FC $FFLAGS -c "$R/tests/$t.F90" 2> err.log || { echo "build_extern: $t failed:" >&2; cat err.log >&2; exit 1; }
  $FC -o "$OUT/bin/$t" $t.o mo_comparisons.o mo_testing_utils.o $FRONT_OBJS shim.o \
      -L"$LIBDIR" -lrte_rrtmgp_hip -Wl,-rpath,'$ORIGIN/../../../rte-rrtmgp_amd' -Wl,-rpath,/opt/rocm/lib
  # the same objects against the reference's own CPU kernels (oracle/_ref/librefkernels.so; the two extension symbols
  # the shim forwards to come from the C oracle): what the programs print when nothing is replaced
  if [ -f "$OUT/librefkernels.so" ]; then
    $FC -o "$OUT/bin/${t}_cpuref" $t.o mo_comparisons.o mo_testing_utils.o $FRONT_OBJS shim.o \
        -L"$OUT" -lrefkernels -L"$HERE" -loracle -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,'$ORIGIN/../..'
  fi
done
# ---- 3. the gas-optics frontend end to end (row f4): oracle/ref_frontend_driver.F90 (ours) = raw table -> k%load ->
#         k%gas_optics -> rte_lw / rte_sw per block of columns, ONE object linked twice (HIP library / reference CPU kernels)
$FC $FFLAGS -c "$HERE/mo_raw_stream.F90" 2> err.log || { cat err.log >&2; exit 1; }
$FC $FFLAGS -c "$HERE/ref_frontend_driver.F90" 2> err.log || { cat err.log >&2; exit 1; }
$FC -o "$OUT/bin/ref_frontend_driver" ref_frontend_driver.o mo_raw_stream.o $FRONT_OBJS shim.o \
    -L"$LIBDIR" -lrte_rrtmgp_hip -Wl,-rpath,'$ORIGIN/../../../rte-rrtmgp_amd' -Wl,-rpath,/opt/rocm/lib
if [ -f "$OUT/librefkernels.so" ]; then
  $FC -o "$OUT/bin/ref_frontend_driver_cpuref" ref_frontend_driver.o mo_raw_stream.o $FRONT_OBJS shim.o \
      -L"$OUT" -lrefkernels -L"$HERE" -loracle -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,'$ORIGIN/../..'
  # ... and with oracle/glue_recorder.c in front of three kernel symbols: records what the reference FRONTEND computed on
  # its way to them (col_gas, tlev, secants, expanded emissivity) -> tests/golden/make_glue_golden.py
  gcc -O1 -fPIC -c "$HERE/glue_recorder.c" -o glue_recorder.o
  $FC -o "$OUT/bin/ref_frontend_driver_glue" ref_frontend_driver.o mo_raw_stream.o $FRONT_OBJS shim.o glue_recorder.o \
      -L"$OUT" -lrefkernels -L"$HERE" -loracle -ldl -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,'$ORIGIN/../..'
fi
# ---- 4. INTEGRATION.md section 4a: the Fortran binding of the factored LW sources (shim/mo_rte_hip_factored.F90) beside the
#         reference's own kernel interface modules, and a data-free program that checks it (oracle/factored_binding_driver.F90)
$FC $FFLAGS -c "$ROOT/shim/mo_rte_hip_factored.F90" 2> err.log || { echo "build_extern: mo_rte_hip_factored.F90 failed:" >&2; cat err.log >&2; exit 1; }
$FC $FFLAGS -c "$HERE/factored_binding_driver.F90" 2> err.log || { echo "build_extern: factored_binding_driver.F90 failed:" >&2; cat err.log >&2; exit 1; }
$FC -o "$OUT/bin/factored_binding_driver" factored_binding_driver.o mo_rte_hip_factored.o \
    -L"$LIBDIR" -lrte_rrtmgp_hip -Wl,-rpath,'$ORIGIN/../../../rte-rrtmgp_amd' -Wl,-rpath,/opt/rocm/lib
# the invariances of the reference's tests/check_equivalence.F90 on the synthetic streams (oracle/ref_equivalence_driver.F90, ours)
$FC $FFLAGS -c "$HERE/ref_equivalence_driver.F90" 2> err.log || { cat err.log >&2; exit 1; }
$FC -o "$OUT/bin/ref_equivalence_driver" ref_equivalence_driver.o mo_raw_stream.o $FRONT_OBJS shim.o \
    -L"$LIBDIR" -lrte_rrtmgp_hip -Wl,-rpath,'$ORIGIN/../../../rte-rrtmgp_amd' -Wl,-rpath,/opt/rocm/lib
if [ -f "$OUT/librefkernels.so" ]; then
  $FC -o "$OUT/bin/ref_equivalence_driver_cpuref" ref_equivalence_driver.o mo_raw_stream.o $FRONT_OBJS shim.o \
      -L"$OUT" -lrefkernels -L"$HERE" -loracle -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,'$ORIGIN/../..'
fi
# the same driver with OpenMP: blocks of columns spread over host threads (concurrent calls on distinct buffers)
$FC $FFLAGS -fopenmp -c "$HERE/ref_frontend_driver.F90" -o ref_frontend_driver_omp.o 2> err.log || { cat err.log >&2; exit 1; }
$FC -fopenmp -o "$OUT/bin/ref_frontend_driver_omp" ref_frontend_driver_omp.o mo_raw_stream.o $FRONT_OBJS shim.o \
    -L"$LIBDIR" -lrte_rrtmgp_hip -Wl,-rpath,'$ORIGIN/../../../rte-rrtmgp_amd' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib
if [ -f "$OUT/librefkernels.so" ]; then
  $FC -fopenmp -o "$OUT/bin/ref_frontend_driver_omp_cpuref" ref_frontend_driver_omp.o mo_raw_stream.o $FRONT_OBJS shim.o \
      -L"$OUT" -lrefkernels -L"$HERE" -loracle -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,'$ORIGIN/../..' -Wl,-rpath,/opt/rocm/lib/llvm/lib
fi
cd "$OUT"; rm -rf "$B"
ls -l "$OUT/bin"
