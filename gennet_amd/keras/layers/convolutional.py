from ...layers import Conv1D, Conv2D, UpSampling1D  # noqa: F401
from .._unused import AveragePooling1D, Conv2DTranspose, MaxPooling1D, MaxPooling2D, UpSampling2D  # noqa: F401
