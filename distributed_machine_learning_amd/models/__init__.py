from .vgg import VGG, VGG11, VGG13, VGG16, VGG19, layer_plan, count_parameters  # noqa: F401
from .summary import summary_string, print_summary  # noqa: F401
from .checkpoint import save_checkpoint, load_checkpoint, expected_keys  # noqa: F401
from .native_vgg import NativeVGG, native_vgg11  # noqa: F401
