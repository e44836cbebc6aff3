cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04o
timeout -k 10 300 python tools/layer_times.py emanet > gpurun_out/r04o/emanet_layers.txt 2> gpurun_out/r04o/emanet_layers.err
timeout -k 10 300 python tools/layer_times.py transunet > gpurun_out/r04o/transunet_layers.txt 2> gpurun_out/r04o/transunet_layers.err
tail -3 gpurun_out/r04o/*.err
head -50 gpurun_out/r04o/emanet_layers.txt
