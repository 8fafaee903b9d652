#!/bin/bash
# A/B of LDS tile shapes for long filters (needs the lab library built with PDWT_TILE_EXPERIMENT=1 in the environment): tools/tilesweep.sh wname rows cols levels
export PDWT_DWT_SPLIT_FWD=0 PDWT_DWT_SPLIT_INV=0 PDWT_NO_PYRAMID=1
for t in ${TILES:-1 2 3 4 5 6}; do
  echo "== $1 $2x$3 L$4 tile $t"
  PDWT_USE_LAB=1 PDWT_FWD_TILE=$t PDWT_INV_TILE2=$t python3 tools/ktimes.py $1 $2 $3 $4 ${5:-1} | awk '{printf "%s %s | ", $1, $3} END{print ""}'
done
