# dispatch order of the team kernel's hot and cold workgroups (SUBZERO_TEAM_RATIO / SUBZERO_TEAM_ORDER / SUBZERO_TEAM_SPLIT): bench headline per setting
for f in 0.15 0.3 0.5 0.7 0.85 1.0; do echo "cold share first $f"; SUBZERO_TEAM_ORDER=2 SUBZERO_TEAM_SPLIT=$f bash profiles/variants.sh gpurun_out/r4_order_s$f.txt base | tail -1; done
