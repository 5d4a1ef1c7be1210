# The drop-in script's call sequence (tools/script_sequence.py) in the four forms of profiles/round6_script_sequence.json.
# Usage: bash tools/r6_seq.sh <tag>
tag=${1:-r6}
for m in host device; do
  timeout -k 10 400 python tools/script_sequence.py --frames 16 --size 5424 --mode $m --out gpurun_out/${tag}_seq_F16_$m.json > gpurun_out/${tag}_seq_F16_$m.log 2>&1 || echo "F16 $m failed"
  tail -1 gpurun_out/${tag}_seq_F16_$m.log | cut -c1-400
  timeout -k 10 300 python tools/script_sequence.py --config C --mode $m --out gpurun_out/${tag}_seq_C_$m.json > gpurun_out/${tag}_seq_C_$m.log 2>&1 || echo "C $m failed"
  tail -1 gpurun_out/${tag}_seq_C_$m.log | cut -c1-400
done
