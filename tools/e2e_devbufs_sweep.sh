# e2e legs of bench.py against the number of device batch buffers per context (MINIVIDEO_DEVBUFS); run on the GPU box
for n in 3 4 5 6 8; do
  MINIVIDEO_DEVBUFS=$n python bench.py --steps 2 --warmup 1 > gpurun_out/r03j_e2e_n$n.json 2> gpurun_out/r03j_e2e_n$n.err || exit 1
  python - $n <<'PY'
import json,sys
b=sys.argv[1]
d=json.loads([l for l in open("gpurun_out/r03j_e2e_n%s.json"%b) if l.startswith("{")][-1])
e=d["end_to_end"]; s=e["stages_rank0"]
print("devbufs",b,"e2e %.3e wall %.3f entropy %.2f d2h %.2f launches %d | rgb-only %.3e | multi %.3e | cli %.3e" % (e["value"], e["wall_s"], s["entropy_decode_host"]["share_of_wall"], s["d2h"]["share_of_wall"], s["kernels"]["launches"], e["rgb_only_rank0"]["value"], d["engine_multi_context"]["value"], d["cli"]["value"]))
PY
done
