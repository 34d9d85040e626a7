#!/bin/bash
# What kind of box is this?  (the overlapped tail of bench.py's inclusive figure is bimodal between boxes of the pool)
echo "host $(hostname) kernel $(uname -r)"
/opt/rocm/bin/rocm-smi --showcomputepartition --showmemorypartition --showclocks --showpower --showperflevel --showfwinfo --showdriverversion 2>&1 | grep -v "^=\|^$" | head -60
/opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|Chip ID|ASIC Revision|Cacheline|SIMDs per CU|Shader Engines|Shader Arrs|Features|Fast F16|Wavefront Size|Max Waves Per CU|uCode" | head -30
