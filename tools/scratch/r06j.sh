cd $GRAFT_REPO_ROOT
bash tools/ab_rounds.sh r05 3 --workload c5 --points 512 --iters 3 --steps 1 --warmup 1
bash tools/ab_rounds.sh r05 3 --workload conv --steps 5 --warmup 1
bash tools/ab_rounds.sh r05 3 --workload c2 --steps 20 --warmup 3
bash tools/ab_rounds.sh r05 2 --workload c5 --points 1024 --iters 10 --steps 1 --warmup 1
