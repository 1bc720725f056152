"""Host-side mirror of the DetMatch-specific mmdet3d layer (SSL detector, 2D+3D wrapper,
OpenPCDet adapter, SSL modules, assigner, runner, optimizer, hooks, box structures) —
the reference's registry/config surface for the hot path."""
