# dev: kNN under different split-K targets of the MFMA plan (GLOC3D_MFMA_WGS: work-groups wanted before K stops splitting)
cd $GRAFT_REPO_ROOT
for cfg in "--n 10000 --q 64" "--n 4541 --q 25" "--n 4541 --q 64" "--n 4541 --q 12" "--n 16000 --q 128" "--n 2000 --q 32"; do
for w in 200 400 800; do echo -n "$cfg wgs $w: "; GLOC3D_MFMA_WGS=$w python tools/bench_knn.py $cfg --reps 100 | tail -1 | python3 -c "
import sys,re; l=sys.stdin.read(); print(re.search(r'([0-9.]+) us/search',l).group(1),'us', re.search(r\"'last_n_tile': (\d+), 'last_k_split': (\d+)\",l).groups())"; done; done
