#!/bin/bash
for n in 22 24 26 28 30; do s=$(date +%s.%N); timeout 900 python tests/big_table_worker.py $n 2>&1 | tail -2; e=$(date +%s.%N); echo "n=$n took $(echo "$e - $s" | bc) s"; done
