cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04ab
timeout -k 10 300 python tools/find_copies.py transunet > gpurun_out/r04ab/copies_transunet.txt 2>&1
head -120 gpurun_out/r04ab/copies_transunet.txt
