#!/bin/bash
# Developer tooling: builds variants of libx266hip.so that differ only in the cache-policy bits of one class of memory instruction, for same-process comparisons
# (tools/probes/gpu_stream_store_policy.py, gpu_dma_load_policy.py):
#   tools/ab_build_policies.sh store      -> tools/_ab/libx266hip_pol<k>.so, k = 0.. : the stream stores (store16_sc1nt / store_tile_sc1nt) as 'sc1 nt' 'sc1' 'nt' '' 'sc0 sc1' 'sc0 sc1 nt'
#   tools/ab_build_policies.sh dma-load   -> tools/_ab/libx266hip_ld<k>.so          : the LDS-DMA loads (global_load_lds_dwordx4 ... nt) as 'nt' '' 'sc1' 'sc0 sc1' 'sc1 nt' 'sc0'
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/_ab
variant() {   # $1 = output name, then sed expressions applied to every file of csrc
  local out=$1; shift
  local T=$(mktemp -d)
  cp -r $R/x266_amd $T/; mkdir -p $T/include; cp $R/include/*.h $T/include/; rm -rf $T/x266_amd/csrc/build $T/x266_amd/*.so
  for e in "$@"; do sed -i "$e" $T/x266_amd/csrc/*.hip $T/x266_amd/csrc/*.hpp; done
  make -C $T/x266_amd/csrc --no-print-directory >/dev/null 2>&1
  cp $T/x266_amd/libx266hip.so $R/tools/_ab/$out
  rm -rf $T
  echo "built tools/_ab/$out"
}
k=0
if [ "$1" = "store" ]; then
  for mods in "sc1 nt" "sc1" "nt" "" "sc0 sc1" "sc0 sc1 nt"; do
    variant libx266hip_pol$k.so "s/%3 sc1 nt\\\\n\\\\tglobal_store_dwordx4 %0, %2, %3 offset:1024 sc1 nt/%3 $mods\\\\n\\\\tglobal_store_dwordx4 %0, %2, %3 offset:1024 $mods/" \
                               "s/global_store_dwordx4 %0, %1, off sc1 nt\\\\n/global_store_dwordx4 %0, %1, off $mods\\\\n/"
    k=$((k+1))
  done
elif [ "$1" = "dma-load" ]; then
  for mods in "nt" "" "sc1" "sc0 sc1" "sc1 nt" "sc0"; do
    variant libx266hip_ld$k.so "s/\\(global_load_lds_dwordx4 %[0-9], %[0-9]\\) nt/\\1 $mods/g"
    k=$((k+1))
  done
elif [ "$1" = "indexed" ]; then   # the offset-table paths of the transform set (plain loads and stores as shipped): 0 shipped, 1 nt loads + sc1 nt stores (also the tile kernel with offsets), 2 stores only, 3 loads only
  variant libx266hip_ix0.so "s/XXXX/XXXX/"
  variant libx266hip_ix1.so "s/load16<!INDEXED>/load16<true>/" "s/store16m<INDEXED ? 0 : 2>/store16m<2>/" "s/tr_tiles_body<INVERSE, false>/tr_tiles_body<INVERSE, true>/"
  variant libx266hip_ix2.so "s/store16m<INDEXED ? 0 : 2>/store16m<2>/"
  variant libx266hip_ix3.so "s/load16<!INDEXED>/load16<true>/"
else
  echo "usage: $0 store | dma-load | indexed"; exit 1
fi
