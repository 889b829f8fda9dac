# same-box A/B of the query-regime encoder chain (tools/enc_chain_profile.py); second line of each pair = round-1 dispatch
for cfg in "minilm 64" "minilm 256" "bge 1" "bge 16" "bge 64" "bge 128" "bge 256"; do set -- $cfg
  for env in "X=0" "CRS_PANEL_KC=384 CRS_ENC_PANEL_MULTI=0 CRS_SPLITK_MAX_TOKENS=2048 CRS_PANEL_MAX_SPLIT=8"; do
    echo -n "$env : "; env $env python3 tools/enc_chain_profile.py $1 $2
  done
done
