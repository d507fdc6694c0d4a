"""`thre3d_atom.thre3d_reprs.constants` of the reference, re-exported from thre3d_atom._keys."""
from thre3d_atom import _keys

_keys.export(globals(),
             {k: v for k, v in _keys.CHECKPOINT_KEYS.items() if k.isupper() and not k.startswith(("CAMERA", "HEMI", "EXTRA"))},
             _keys.STATE_DICT_NAMES)
