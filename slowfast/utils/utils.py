from csts_amd.losses import frame_softmax, sim_matrix  # noqa: F401
