cd /root/repo
for cfg in "128 0" "128 256" "128 512" "136 0" "130 0" "129 0"; do
  set -- $cfg
  echo "=== DBG $1 tiles $2"
  PH_DBG=$1 PH_TILES=$2 python tools/mb_tile_phases.py 2>&1 | grep -v amdgpu.ids
done
