#!/bin/bash
for d in 0 1 2 3 4; do echo "DBG=$d"; BSVI_X6_DEBUG=$d python3 tools/r4/x6_probe.py 2>&1 | grep "^M" | cut -c1-110; done
bash tools/r4/cfg5_x6_ab.sh
