for v in "GSPN_PREAGG_U=4" "GSPN_PREAGG_U=2" "GSPN_PREAGG_U=1" "GSPN_PREAGG_U=2 GSPN_PREAGG_BLOCKS=4096" "GSPN_PREAGG_U=1 GSPN_PREAGG_BLOCKS=8192" "GSPN_PREAGG_U=4 GSPN_PREAGG_BLOCKS=1024"; do
  echo "== $v"; env $v bash tools/trace_layers.sh > /dev/null; grep preagg_fwd gpurun_out/layers_timeline.txt | awk '{print $4, $5}' | tr '\n' ' '; echo
done
