# bwd time of attention shapes for several builds of the library: bash tools/scripts/attn_variants.sh <shape> <variant names...>  (main = in-tree)
S=$1; shift
for v in "$@"; do
  if [ $v = main ]; then r=$(python tools/attn_bench.py --only $S 2>/dev/null | tail -1); else r=$(CSTS_HIP_LIB=tools/diag/libcsts_hip_$v.so python tools/attn_bench.py --only $S 2>/dev/null | tail -1); fi
  echo "$v: $r"
done
