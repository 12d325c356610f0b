"""TEST-ONLY stand-in for the handful of pydrake names the LeafSystem adapter touches (pydrake is not in this
image).  Written from Drake's documented Python API, not from Drake sources; only what
quadruped_drake_amd.controller.make_leaf_system calls exists here."""
