"""Names that bbhMahoGANy.py IMPORTS from keras (:33-43) but never instantiates on the BBH path.  They exist so that the script's
import lines work unchanged against gennet_amd.keras; constructing one raises, instead of silently running something else."""


def _placeholder(name, where):
    def __init__(self, *a, **k):
        raise NotImplementedError('%s is imported by bbhMahoGANy.py (%s) but never used on the BBH hot path; gennet_amd does not '
                                  'provide it' % (name, where))
    return type(name, (object,), {'__init__': __init__, '__doc__': 'placeholder for keras %s (not on the hot path)' % name})


GlobalAveragePooling1D = _placeholder('GlobalAveragePooling1D', 'bbhMahoGANy.py:33')
AlphaDropout = _placeholder('AlphaDropout', 'bbhMahoGANy.py:34')
GaussianDropout = _placeholder('GaussianDropout', 'bbhMahoGANy.py:34')
GaussianNoise = _placeholder('GaussianNoise', 'bbhMahoGANy.py:34')
UpSampling2D = _placeholder('UpSampling2D', 'bbhMahoGANy.py:37')
Conv2DTranspose = _placeholder('Conv2DTranspose', 'bbhMahoGANy.py:37')
MaxPooling2D = _placeholder('MaxPooling2D', 'bbhMahoGANy.py:38')
AveragePooling1D = _placeholder('AveragePooling1D', 'bbhMahoGANy.py:38')
MaxPooling1D = _placeholder('MaxPooling1D', 'bbhMahoGANy.py:38')
ThresholdedReLU = _placeholder('ThresholdedReLU', 'bbhMahoGANy.py:39')
RMSprop = _placeholder('RMSprop', 'bbhMahoGANy.py:43')
Adagrad = _placeholder('Adagrad', 'bbhMahoGANy.py:43')
Adadelta = _placeholder('Adadelta', 'bbhMahoGANy.py:43')
Adamax = _placeholder('Adamax', 'bbhMahoGANy.py:43')
Nadam = _placeholder('Nadam', 'bbhMahoGANy.py:43')
