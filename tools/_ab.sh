for nw in 0 1; do
  echo "== NW4=$nw"
  BESO_FUSED_LEVEL_MAX=1 BESO_FUSED_NW4=$nw python tools/phase_stamps.py 4096 2>&1 | grep -v amdgpu.ids
done
