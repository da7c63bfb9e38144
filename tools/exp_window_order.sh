#!/bin/bash
# Run ON THE GPU BOX (needs profiles/r06_window_by_area.patch applied): numbering of the wide nodes for the LDS window -- breadth-first against by surface area:
# build/lib_bfs.so, build/lib_area.so (tools/build_variant.sh ... -DCL2_WINDOW_BY_AREA=0 / 1), alternating, 8 sample streams.
for scene in glass blob interior; do
  for lib in bfs area bfs area; do
    echo "== $scene $lib"
    CL2_LIB=build/lib_$lib.so python tools/exp_mesh_flags_ab.py $scene 8 0 2>&1 | grep flags
  done
done
# window size on top of the by-area numbering (debug bits 20-23: units of 32 nodes; default 32 cache-resident / 64 streaming)
for scene in glass blob interior; do
  echo "== $scene area, windows default / 32 / 64 / 96 / 128 / 192"
  CL2_LIB=build/lib_area.so python tools/exp_mesh_flags_ab.py $scene 8 0 0x100000 0x200000 0x300000 0x400000 0x600000 2>&1 | grep flags
done
