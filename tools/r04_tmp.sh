for lib in libgspn_hip.so libgspn_hip_csru16.so; do echo "== $lib"; GSPN_HIP_LIB=$GRAFT_REPO_ROOT/gspn_amd/lib/$lib python tools/r04_gather_family.py 2>&1 | grep "csr"; done
