from ...layers import Conv1D, Conv2D, UpSampling1D  # noqa: F401
