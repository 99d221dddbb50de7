export RANK_SIM_FIRST=1
for ns in 3 5 8 15; do for tg in 512 2048; do echo "streams=$ns target=$tg: $(NMFK_MERGE=0 NMFK_STREAMS=$ns NMFK_TARGET_WGS=$tg timeout 200 python scripts/rank_sim.py 8 2>&1 | tail -1 | cut -c45-75)"; done; done
