"""Import shim for `jaxtyping` (test infrastructure only, see oracle/README.md).

The reference only uses `Float[Tensor, "..."]` as an annotation
(reflect_sampling_nerf_field.py:7, reflect_sampling_nerf_components.py:6), so a
subscriptable no-op is enough.
"""


class _Subscriptable:
    def __class_getitem__(cls, item):
        return cls

    def __getitem__(self, item):
        return self


class Float(_Subscriptable):
    pass


class Int(_Subscriptable):
    pass


class Shaped(_Subscriptable):
    pass


class Bool(_Subscriptable):
    pass
