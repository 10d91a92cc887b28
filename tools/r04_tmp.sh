python -m pytest tests -m gpu -q 2>&1 | tail -3
for v in 0 1; do echo "GSPN_FUSE_POOLN=$v"; GSPN_FUSE_POOLN=$v python tools/c3_leg.py 8; done
