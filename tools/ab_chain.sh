# same-box A/B of the query-regime encoder chain (tools/enc_chain_profile.py); CRS_PANEL_KC=384 CRS_ENC_PANEL_MULTI=0 = round-1 staging / dispatch
for cfg in "minilm 64" "bge 64" "bge 16" "bge 256" "bge 1"; do set -- $cfg
  for env in "X=0" "CRS_PANEL_KC=384 CRS_ENC_PANEL_MULTI=0"; do
    echo -n "$env : "; env $env python3 tools/enc_chain_profile.py $1 $2
  done
done
