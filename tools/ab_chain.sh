# same-box A/B of the query-regime encoder chain (tools/enc_chain_profile.py)
for cfg in "minilm 64" "bge 64" "bge 256"; do set -- $cfg
  for env in "X=0" "CRS_PANEL_TM=64" "CRS_PANEL_TM=64 CRS_PANEL_KC=128" "CRS_PANEL_TM=128 CRS_PANEL_KC=128"; do
    echo -n "$env : "; env $env python3 tools/enc_chain_profile.py $1 $2
  done
done
