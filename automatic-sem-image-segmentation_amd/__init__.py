"""MI355X-native CycleGAN -> MultiResUNet training hot path (drop-in for the Keras-3 torch backend path of
BAMresearch/automatic-sem-image-segmentation, Releases/Version 1.2.0).

The directory name carries the reference's name and is therefore not a valid Python identifier; import it with
``importlib.import_module("automatic-sem-image-segmentation_amd")`` or put this directory on ``sys.path`` and
``import CycleGAN, UNet_Segmentation`` exactly like the reference's ``StartProcess.py`` does.
"""
__version__ = "0.1.0"
