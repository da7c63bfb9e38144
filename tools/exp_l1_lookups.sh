#!/bin/bash
# Run ON THE GPU BOX: what does one more L1 look-up per wide-node visit cost?  Builds base / extra1 / extra2 of
# tools/build_variant.sh (-DCL2_EXTRA_NODE_LOADS=N: N more 16-byte loads of the node's own line per visit), alternating,
# on configs 3 and 5 with 8 sample streams.  If the launch time grows with the look-ups, the L1's look-up rate binds the walk.
for scene in glass interior; do
  for lib in base extra1 extra2 extra4 base extra1 extra2 extra4; do
    echo "== $scene $lib"
    CL2_LIB=build/lib_$lib.so python tools/exp_mesh_flags_ab.py $scene 8 0 2>&1 | grep flags
  done
done
