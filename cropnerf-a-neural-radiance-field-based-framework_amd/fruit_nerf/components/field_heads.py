"""``SemanticFieldHead`` -- ``crop_nerf/fruit_nerf/components/field_heads.py:29-40``: a bias-ful ``Linear(in_dim,
num_classes)`` without activation.  On the HIP path it is the ``sem_head_weight/bias`` pair of ``cn_field_params``
(folded into the preceding Linear by the fused kernel); this class only names the parameters."""


class SemanticFieldHead:
    def __init__(self, in_dim: int, num_classes: int, activation=None) -> None:
        assert activation is None
        self.in_dim = in_dim
        self.num_classes = num_classes

    weight_key = "field.field_head_semantics.net.weight"
    bias_key = "field.field_head_semantics.net.bias"
