cd $GRAFT_REPO_ROOT
for rep in 1 2; do bash tools/chunk_sweep.sh c2 4 5 6 7 8 10 -- --steps 20 --warmup 3; done 2>&1 | tee gpurun_out/r06m_chunk_sweep_c2.txt
