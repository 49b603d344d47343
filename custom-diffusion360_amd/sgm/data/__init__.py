"""Camera conventions of the reference's CO3D data module (sgm/data/data_co3d.py).  Only the camera arithmetic either side of
the pose path is mirrored; the image dataset / Lightning data module are I/O and out of scope."""
