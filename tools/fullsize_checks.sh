#!/bin/bash
# Full-size exactness checks of the read kernels: every bench configuration, single-value and OneOf predicates.
# Stops at the first failing step.
set -e
cd "$(dirname "$0")/.."
for c in cfg1 cfg2 cfg3 cfg4 cfg5sin cfg5mix; do
  timeout -k 10 300 python3 -u tools/exact_check.py $c ${PASSES:-9}
done
for n in 2 3 5 9 16; do
  timeout -k 10 300 python3 -u tools/oneof_check.py cfg4 $n 3
  timeout -k 10 300 python3 -u tools/oneof_check.py cfg1 $n 3
done
timeout -k 10 300 python3 -u tools/oneof_check.py cfg3 3 3
echo "all full-size checks passed"
