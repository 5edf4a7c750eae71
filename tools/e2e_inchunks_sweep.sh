# e2e legs of bench.py against the number of page-locked input chunks (MINIVIDEO_IN_CHUNKS); run on the GPU box
for n in 0 6 8 12 16; do
  MINIVIDEO_IN_CHUNKS=$n python bench.py --steps 2 --warmup 1 > gpurun_out/r03j_e2e_c$n.json 2> gpurun_out/r03j_e2e_c$n.err || exit 1
  python - $n <<'PY'
import json,sys
b=sys.argv[1]
d=json.loads([l for l in open("gpurun_out/r03j_e2e_c%s.json"%b) if l.startswith("{")][-1])
e=d["end_to_end"]; s=e["stages_rank0"]
print("in_chunks",b,"e2e %.3e wall %.3f entropy %.2f h2d %.2f (%.1f GB/s) d2h %.2f | rgb-only %.3e | multi %.3e | cli %.3e cold %.3f" % (e["value"], e["wall_s"], s["entropy_decode_host"]["share_of_wall"], s["h2d"]["share_of_wall"], s["h2d"]["GB/s"], s["d2h"]["share_of_wall"], e["rgb_only_rank0"]["value"], d["engine_multi_context"]["value"], d["cli"]["value"], e["cold_call_s"]))
PY
done
