python tools/r04_tmp.py 2>&1 | grep "FP3 lists"
for T in 0 192 256 384 512 768; do echo "== GSPN_CSR_LONG=$T"; GSPN_GATHER_KIND=S GSPN_CSR_LONG=$T python tools/r04_gather_family.py 2>&1 | grep "FP3 pre"; done
