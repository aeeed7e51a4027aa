cd /root/repo
for pad in 0 20000 40000; do
  echo "LDSPAD=$pad"; CG3D_TILE_INFO=1 CG3D_TILE_LDSPAD=$pad python tools/mb_tile_one.py 4 128 128 2>&1 | grep -v amdgpu.ids
done
for u in 255 383; do echo "UCAP=$u"; CG3D_TILE_INFO=1 CG3D_TILE_UCAP=$u python tools/mb_tile_one.py 4 128 128 2>&1 | grep -v amdgpu.ids; done
