for b in 0 64 128 256 512; do
  python bench.py --steps 2 --warmup 1 --e2e-batch $b > gpurun_out/r03j_e2e_b$b.json 2> gpurun_out/r03j_e2e_b$b.err || exit 1
  python - $b <<'PY'
import json,sys
b=sys.argv[1]
d=json.loads([l for l in open("gpurun_out/r03j_e2e_b%s.json"%b) if l.startswith("{")][-1])
e=d["end_to_end"]; s=e["stages_rank0"]
print("batch",b,"e2e %.3e wall %.3f entropy %.2f d2h %.2f launches %d | rgb-only %.3e | multi %.3e" % (e["value"], e["wall_s"], s["entropy_decode_host"]["share_of_wall"], s["d2h"]["share_of_wall"], s["kernels"]["launches"], e["rgb_only_rank0"]["value"], d["engine_multi_context"]["value"]))
PY
done
