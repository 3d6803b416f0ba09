"""Pieces of bench.py: the timed step lives in bench.py, the lines around it (ray-batch loader, secondary lines, CPU baseline + parity leg,
roofline arithmetic and the compact driver line) live here.  Only cpu_baseline.py may touch oracle/ (it is the checker and the CPU baseline)."""
