#!/bin/sh
# The boundary-condition block of the reference's RFMIP-SW driver as a callable (VERDICT r5
# "missing" 5): the statements are cut out of examples/rfmip-clear-sky/rrtmgp_rfmip_sw.F90 where it lies, by the block's own
# comment lines, into oracle/_ref/rfmip_build/*.inc (git-ignored, never copied into the repository), and compiled inside
# oracle/rfmip_sw_glue_wrapper.F90 -> oracle/_ref/librfmipswglue.so.  Fails if the reference file no longer has the markers.
set -e
R=${REFERENCE_ROOT:-/root/reference}
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
B=$OUT/rfmip_build
SRC=$R/examples/rfmip-clear-sky/rrtmgp_rfmip_sw.F90
rm -rf "$B"; mkdir -p "$B"
# from "What's the total solar irradiance assumed by RRTMGP?" up to (not including) the rte_sw call's comment
sed -n "/What's the total solar irradiance assumed by RRTMGP/,/and compute the spectrally-resolved fluxes/p" "$SRC" | sed '$d' > "$B/rfmip_sw_block.inc"
# from "Zero out fluxes" to the end of that loop (the first "end do" at loop level: four spaces)
sed -n '/Zero out fluxes for which the original solar zenith angle/,/^    end do/p' "$SRC" > "$B/rfmip_sw_mask.inc"
grep -E '^ *real\(wp\), *parameter *:: *deg_to_rad' "$SRC" > "$B/rfmip_sw_param.inc"
for f in rfmip_sw_block.inc rfmip_sw_mask.inc rfmip_sw_param.inc; do
  [ -s "$B/$f" ] || { echo "build_rfmip_sw_glue: marker for $f not found in $SRC" >&2; exit 1; }
done
grep -q "total_solar_irradiance(icol,b)/def_tsi(icol)" "$B/rfmip_sw_block.inc" || { echo "build_rfmip_sw_glue: unexpected block" >&2; exit 1; }
grep -q "usecol(icol,b)" "$B/rfmip_sw_mask.inc" || { echo "build_rfmip_sw_glue: unexpected mask block" >&2; exit 1; }
cd "$B"
$FC -O2 -fPIC -c "$R/rte/kernels/mo_rte_kind.F90"
$FC -O2 -fPIC -cpp -I"$B" -c "$HERE/rfmip_sw_glue_wrapper.F90" -o wrapper.o
$FC -shared -o "$OUT/librfmipswglue.so" wrapper.o mo_rte_kind.o
cd "$OUT"; rm -rf "$B"
ls -l "$OUT/librfmipswglue.so"
