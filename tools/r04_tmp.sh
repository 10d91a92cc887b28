for v in "" _nmq1 _nmq4; do echo "== libgspn_hip$v.so"; GSPN_HIP_LIB=$GRAFT_REPO_ROOT/gspn_amd/lib/libgspn_hip$v.so python tools/r04_nm.py 2>&1 | grep nn_distance; done
python -m pytest tests -m gpu -q -k "nn_distance or policy or chamfer" 2>&1 | tail -2
