mkdir -p gpurun_out/r03a
timeout 1500 python3 -m pytest tests -q -x -m gpu > gpurun_out/r03a/pytest.log 2>&1; tail -3 gpurun_out/r03a/pytest.log
timeout 300 python3 bench.py > gpurun_out/r03a/bench_default.json 2> gpurun_out/r03a/bench_default.err
timeout 300 python3 bench.py --config cfg5 --no-extras --no-cpu-baseline > gpurun_out/r03a/bench_cfg5.json 2> gpurun_out/r03a/bench_cfg5.err
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 timeout 300 python3 bench.py --force-dist --dist-backend nccl --no-extras --no-cpu-baseline > gpurun_out/r03a/bench_nccl_world1.json 2> gpurun_out/r03a/bench_nccl_world1.err; echo "nccl exit $?"
export TMPDIR=/tmp
tools/prof.sh r03a_cfg2_b16 --config cfg2 --batch 16 > gpurun_out/r03a/prof_b16.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/prof_r03a_cfg2_b16 gpurun_out/r03a/traffic_cfg2_b16.json cfg2 > gpurun_out/r03a/rocprofv3_summary_cfg2_b16.txt 2>&1
rm -rf gpurun_out/prof_r03a_cfg2_b16/*/*/*.db 2>/dev/null
du -sh gpurun_out/prof_r03a_cfg2_b16
echo done
