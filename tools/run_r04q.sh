cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04q
for shape in "3072 768 gelu" "2304 768 bias" "3072 768 none"; do
  SIS_HIP_LIB=libsis_hip_trace.so timeout -k 10 200 python tools/trace_gemm256.py $shape > "gpurun_out/r04q/trace_$(echo $shape | tr ' ' '_').txt" 2>&1
  cat "gpurun_out/r04q/trace_$(echo $shape | tr ' ' '_').txt"
done
