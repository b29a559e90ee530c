#!/bin/bash
# tools/ab_solo.sh OPS ROUNDS lib.so ... -- like tools/ab_bench.py but every library in a process of its own, the processes
# alternating on one box: a library loaded second into one process allocates its scratch behind the first one's and its
# table / scratch placement alone moves sign by 15 % and verify by 3 % (profiles/r04_ab_prio.txt), which hides small effects.
OPS=$1; ROUNDS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    python tools/ab_bench.py $lib --ops $OPS --rounds 3 2>&1 | grep -E "^(x25519|sign|verify|keypair) " | sed "s/^/round $r  /"
  done
done
