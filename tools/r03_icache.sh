set -x
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_icache
cd /tmp
rocprofv3 -L > $REPO/gpurun_out/r03_icache/counters.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" $REPO/gpurun_out/r03_icache/counters.txt | sort -u > $REPO/gpurun_out/r03_icache/icache_names.txt
cat $REPO/gpurun_out/r03_icache/icache_names.txt
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $REPO/gpurun_out/r03_icache/g1 -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --model models/biped/biped.xml > $REPO/gpurun_out/r03_icache/g1.log 2>&1
echo rc=$?
tail -2 $REPO/gpurun_out/r03_icache/g1.log
cd $REPO
python3 tools/pmc_summary.py gpurun_out/r03_icache 2>&1 | tail -20
python tools/ab_bench.py ambersim_amd/csrc/libamx.so 2>&1 | tail -3
