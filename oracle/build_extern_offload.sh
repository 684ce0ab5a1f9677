#!/bin/sh
# The reference's UNCHANGED Fortran frontend in extern mode, compiled with OpenMP target offload for gfx950
# (flang -fopenmp --offload-arch=gfx950): the frontend's own `!$omp target data` / `!$omp target enter data` regions
# (rte/frontend/mo_rte_lw.F90:327-365,443-449; rrtmgp/frontend/mo_gas_optics_rrtmgp.F90:594-609,893-912) then keep its arrays
# on the device and run its glue loops there, and the kernel symbols of librte_rrtmgp_hip.so receive HOST addresses of
# MAPPED arrays, which the library resolves with omp_get_mapped_ptr (csrc/runtime.hip, RTE_HIP_OMP_MAPPED).
# Output: oracle/_ref/bin/ref_frontend_driver_offload (binary only, git-ignored, travels to the GPU box).
# Test infrastructure; needs /root/reference + flang.
set -e
R=${REFERENCE_ROOT:-/root/reference}
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
LIBDIR=$ROOT/rte-rrtmgp_amd
OUT=$HERE/_ref
B=$OUT/offload_build
rm -rf "$B"; mkdir -p "$B" "$OUT/bin"
FFLAGS="${OFFLOAD_EXTRA:-} -O2 -fPIC -DAMDFLANG_WORKAROUND -fopenmp --offload-arch=gfx950"
API="rte/kernels/mo_rte_kind.F90 rte/kernels/api/mo_rte_util_array.F90 rte/kernels/mo_gas_optics_constants.F90 \
rte/kernels/api/mo_fluxes_broadband_kernels.F90 rte/kernels/api/mo_gas_optics_utils.F90 \
rte/kernels/api/mo_optical_props_kernels.F90 rte/kernels/api/mo_rte_solver_kernels.F90 \
rrtmgp/kernels/api/mo_gas_optics_rrtmgp_kernels.F90 rrtmgp/kernels/api/mo_cloud_optics_rrtmgp_kernels.F90"
FRONT="rte/frontend/mo_rte_config.F90 rte/frontend/mo_rte_util_array_validation.F90 rte/frontend/mo_optical_props.F90 \
rte/frontend/mo_source_functions.F90 rte/frontend/mo_fluxes.F90 rte/frontend/mo_rte_lw.F90 rte/frontend/mo_rte_sw.F90 \
rte/frontend/gas-optics-template/mo_gas_optics_util_string.F90 rte/frontend/gas-optics-template/mo_gas_concentrations.F90 \
rte/frontend/gas-optics-template/mo_gas_optics.F90 \
rrtmgp/frontend/mo_gas_optics_rrtmgp.F90 rrtmgp/frontend/mo_cloud_optics_rrtmgp.F90 rte/extensions/mo_fluxes_byband.F90"
cd "$B"
for f in $API $FRONT; do
  $FC $FFLAGS -c "$R/$f" 2> err.log || { echo "build_extern_offload: $f failed:" >&2; tail -30 err.log >&2; exit 1; }
done
FRONT_OBJS=$(for f in $API $FRONT; do echo "$(basename "$f" .F90).o"; done)
$FC $FFLAGS -c "$ROOT/shim/rte_hip_fortran_shim.F90" -o shim.o 2> err.log || { cat err.log >&2; exit 1; }
$FC $FFLAGS -c "$HERE/mo_raw_stream.F90" 2> err.log || { cat err.log >&2; exit 1; }
# (the driver itself without its own OpenMP threading: -fopenmp would make its `!$omp parallel` block loop live; one host
#  thread is what this build is about, so the sentinel lines stay comments)
$FC -O2 -fPIC -DAMDFLANG_WORKAROUND -c "$HERE/ref_frontend_driver.F90" 2> err.log || { cat err.log >&2; exit 1; }
$FC -fopenmp --offload-arch=gfx950 -o "$OUT/bin/ref_frontend_driver_offload" ref_frontend_driver.o mo_raw_stream.o $FRONT_OBJS shim.o \
    -L"$LIBDIR" -lrte_rrtmgp_hip -Wl,-rpath,'$ORIGIN/../../../rte-rrtmgp_amd' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib \
    2> err.log || { echo "build_extern_offload: link failed:" >&2; tail -30 err.log >&2; exit 1; }
# the mechanism on its own: a C host program with a `target data` region calling a kernel symbol with mapped host addresses
/opt/rocm/lib/llvm/bin/clang -O2 -fopenmp --offload-arch=gfx950 -I"$ROOT/include" "$HERE/omp_mapped_check.c" -o "$OUT/bin/omp_mapped_check" \
    -L"$LIBDIR" -lrte_rrtmgp_hip -lm -Wl,-rpath,'$ORIGIN/../../../rte-rrtmgp_amd' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib \
    2> err.log || { echo "build_extern_offload: omp_mapped_check failed:" >&2; tail -30 err.log >&2; exit 1; }
cd "$OUT"; rm -rf "$B"
ls -l "$OUT/bin/ref_frontend_driver_offload" "$OUT/bin/omp_mapped_check"
