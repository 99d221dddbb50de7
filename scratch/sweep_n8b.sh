export RANK_SIM_FIRST=1
for tg in 64 128 240 512; do echo "merge=0 target=$tg: $(NMFK_MERGE=0 NMFK_TARGET_WGS=$tg timeout 200 python scripts/rank_sim.py 8 2>&1 | tail -1 | cut -c45-75)"; done
