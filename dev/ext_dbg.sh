#!/bin/bash
for nt in 256 512 1024; do echo "== nt=$nt"; for a in "9800 1500 0x40" "9800 150001 0x0" "9800 150001 0x08"; do PGA_WIDE_NT=$nt timeout 90 python dev/ext_one.py $a 2>&1 | grep "ms wall"; done; done
