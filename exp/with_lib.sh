#!/bin/bash
# run a command with another build of the engine in the product's place, and put the shipped one back WHATEVER happens:
#   bash exp/with_lib.sh exp/_dbg/libh2e_dbg.so -- python bench.py ...
# (the product has no library switch by environment; a debug / sanitised build left behind as halo2ecc_s_amd/libh2e.so would be
# what every later test and bench line runs - so the swap is undone under a trap, and the backup sits beside the library under a
# name of its own)
ALT=$1; shift; [ "$1" = "--" ] && shift
cd "$(dirname "$0")/.."
LIB=halo2ecc_s_amd/libh2e.so
KEEP=halo2ecc_s_amd/.libh2e_shipped.$$.so
[ -f "$ALT" ] || { echo "with_lib: $ALT not found" >&2; exit 2; }
cp "$LIB" "$KEEP" || exit 2
restore() { [ -f "$KEEP" ] && mv -f "$KEEP" "$LIB"; }
trap restore EXIT
trap 'restore; exit 130' INT TERM HUP
cp "$ALT" "$LIB"
"$@"
