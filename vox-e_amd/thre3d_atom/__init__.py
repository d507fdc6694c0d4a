"""`thre3d_atom` -- host-side mirror of the Vox-E rendering/optimisation API on top of the MI355X HIP
renderer (voxe_hip / libvoxe_hip.so).

The module paths and public names are those of TAU-VAILab/Vox-E's `thre3d_atom` package so that
callers are source compatible and reference checkpoints (which pickle
`thre3d_atom.thre3d_reprs.renderers.render_sh_voxel_grid` etc. by qualified name) unpickle.
The implementation is new: the sampler -> point-processor -> accumulator chain and its autograd
graph are one fused HIP forward kernel and one fused HIP backward kernel behind a C ABI.
"""
