# session 2, call 3: where the step goes at other utterance lengths (timelines at 2 s, 5 s, 11 s)
mkdir -p gpurun_out/s2c3
for S in 2 5 11; do
  bash scripts/gpu_timeline.sh s2c3_tl$S --seconds $S > /dev/null 2>&1
  cp gpurun_out/s2c3_tl$S/step_timeline.txt gpurun_out/s2c3/step_timeline_fp32_${S}s.txt
  tail -1 gpurun_out/s2c3/step_timeline_fp32_${S}s.txt
done
