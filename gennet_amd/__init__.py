"""gennet_amd: MI355X-native (gfx950) implementation of GenNet's BBH training hot path.

Python host code mirrors the Keras-style surface used by BBH_version/bbhMahoGANy.py and the template
synthesiser surface of BBH_version/gw_template_maker.py; all arithmetic runs in hand-written HIP kernels
behind the C ABI declared in include/gennet_hip.h (libgennet_hip.so).
"""
__version__ = '0.1.0'
