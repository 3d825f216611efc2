"""`from pt_utils import MaskedQueryAndGroup, MaskedMaxPool, MaskedUpsample` (reference
local_aggregation_operators.py:13, resnet.py:11, segmentation_head.py:12) resolves here when
`drop_in/` precedes the reference's `ops/pt_custom_ops` on sys.path."""
from closerlook3d_amd.pt_utils import *  # noqa: F401,F403
from closerlook3d_amd.pt_utils import (GroupingOperation, MaskedGridSubsampling, MaskedMaxPool,  # noqa: F401
                                       MaskedNearestQuery, MaskedNearestQueryAndGroup,
                                       MaskedOrderedBallQuery, MaskedQueryAndGroup, MaskedUpsample,
                                       ball_query_cache, grouping_operation, join_index_stream,
                                       masked_grid_subsampling, masked_nearest_query, masked_ordered_ball_query,
                                       prefetch_geometry)
