export RANK_SIM_FIRST=1
for c in 0 2 1; do echo "chain=$c: $(NMFK_CHAIN=$c NMFK_HOST_TIMING=1 timeout 200 python scripts/rank_sim.py 8 2>&1 | tail -2 | cut -c1-140 | tr '\n' ' ')"; done
for g in 4 6; do echo "chain=1 G=$g: $(NMFK_CHAIN=1 NMFK_CHAIN_G=$g timeout 200 python scripts/rank_sim.py 8 2>&1 | tail -1 | cut -c45-75)"; done
echo "N=4 chain=1: $(NMFK_CHAIN=1 timeout 200 python scripts/rank_sim.py 4 2>&1 | tail -1 | cut -c45-75)"
