# session 2, call 5: A/B of the half-tile row split (same box): sweep at the affected lengths, bucketed training, bucketed inference
mkdir -p gpurun_out/s2c5
timeout 900 python -m pytest tests/test_gpu_config2.py -q -m gpu -x 2>&1 | tail -3
for V in 0 1 0 1; do
  echo "no_row_split=$V" | tee -a gpurun_out/s2c5/ab.txt
  ST_TUNE=no_row_split=$V python scripts/bench_varlen_train.py --sweep 601 701 1001 1101 1201 --out gpurun_out/s2c5/sweep_$V.json 2>/dev/null | tail -5 | cut -c1-110 | tee -a gpurun_out/s2c5/ab.txt
done
for V in 0 1 0 1; do
  echo "no_row_split=$V" | tee -a gpurun_out/s2c5/ab.txt
  ST_TUNE=no_row_split=$V python scripts/bench_varlen_train.py --batch 32 --mels 80 --orders bucketed --out gpurun_out/s2c5/varlen_$V.json 2>/dev/null | tail -1 | cut -c1-260 | tee -a gpurun_out/s2c5/ab.txt
  ST_TUNE=no_row_split=$V timeout 600 python scripts/bench_inference.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(v if not isinstance(v,(dict,list)) else '') for k,v in d.items() if 'utt' in k or 'per_s' in k}); print(json.dumps(d)[:900])" | tee -a gpurun_out/s2c5/ab.txt
done
