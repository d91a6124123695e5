#!/bin/bash
# A/B build of the library with ONE source compiled with extra flags: bash tools/scripts/build_variant.sh <name> <source stem> <flags...>
# -> tools/diag/libcsts_hip_<name>.so (all other objects taken from csts_amd/csrc/build; run `make -C csts_amd/csrc` first).
# With STAMPS=1 the diagnostics (in-kernel stamps) form of that source is built (-DCSTS_ATTN_STAMPS for the attention parts).
set -euo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
NAME=$1; STEM=$2; shift 2
cd "$R/csts_amd/csrc"
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-pass-failed -Wno-inline-asm"
PER=$(sed -n "s/^FLAGS_${STEM} *:= *//p" Makefile | head -1)
EXTRA=""
if [ "${STAMPS:-0}" = 1 ]; then EXTRA="-DCSTS_ATTN_STAMPS"; fi
mkdir -p build/var ../../tools/diag
hipcc $BASE $PER $EXTRA "$@" -c ${STEM}.hip -o build/var/${STEM}_${NAME}.o
SRCS=$(sed -n "s/^SRCS *:= *//p" Makefile | head -1)
OBJS=$(for f in $SRCS; do o=build/${f%.hip}.o; [ "$o" = "build/${STEM}.o" ] || echo $o; done)
if [ "${STAMPS:-0}" = 1 ] && [ "$STEM" = attention_dkv ]; then
  hipcc $BASE -DCSTS_ATTN_STAMPS -c attention.hip -o build/var/attention_stampsbase.o
  OBJS=$(echo $OBJS | tr ' ' '\n' | grep -v "build/attention.o" | tr '\n' ' ')" build/var/attention_stampsbase.o"
fi
[ -n "$OBJS" ] || { echo "no objects"; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/diag/libcsts_hip_${NAME}.so $OBJS build/var/${STEM}_${NAME}.o
nm -D ../../tools/diag/libcsts_hip_${NAME}.so | grep -q csts_last_error || { echo "incomplete library"; exit 1; }
echo "built tools/diag/libcsts_hip_${NAME}.so ($STEM: $PER $EXTRA $*)"
