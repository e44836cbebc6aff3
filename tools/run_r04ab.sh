cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04ab
timeout -k 10 300 python tools/find_copies.py emanet > gpurun_out/r04ab/copies_emanet.txt 2>&1
grep -v "^   \s" gpurun_out/r04ab/copies_emanet.txt | head -60
