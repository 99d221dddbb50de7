for c in 0 2 1; do echo "k=2..7 R=4 chain=$c: $(NMFK_CHAIN=$c timeout 100 python scripts/microbench.py 300 2 7 4 | cut -c26-60)"; done
for c in 0 1; do echo "k=2..7 R=8 chain=$c: $(NMFK_CHAIN=$c timeout 100 python scripts/microbench.py 300 2 7 8 | cut -c26-60)"; done
