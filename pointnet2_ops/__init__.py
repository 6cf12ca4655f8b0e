"""Drop-in for the third-party `pointnet2_ops` package the reference imports
(ptt/models/backbones_3d/pointnet2/pointnet2_utils.py:24: `import pointnet2_ops._ext as _ext`).
Putting this repo on PYTHONPATH makes the reference's own wrappers run on the MI355X kernels."""
