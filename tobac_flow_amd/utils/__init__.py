"""Utils: helpers that do not depend on other modules of the package
(reference: tobac_flow/utils/__init__.py; only the hot path's helpers exist here)."""
from tobac_flow_amd.utils.datetime_utils import *  # noqa: F401,F403
from tobac_flow_amd.utils.flow_utils import *  # noqa: F401,F403
from tobac_flow_amd.utils.label_utils import *  # noqa: F401,F403
from tobac_flow_amd.utils.normalisation_utils import *  # noqa: F401,F403
from tobac_flow_amd.utils.stats_utils import *  # noqa: F401,F403
