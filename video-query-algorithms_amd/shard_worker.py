"""One non-root rank of a served :class:`ShardedFeatureDB`: started once per further GPU by ``ShardedFeatureDB.open`` (as a
fresh process, before the broker's process touches its GPU), holds its row range of the feature store on its own MI355X and
executes what rank 0 announces -- scans, partitions, gathers -- until the database is closed."""
import os
import sys

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from video_query_algorithms_amd.sharded_db import worker_main
else:
    from .sharded_db import worker_main

if __name__ == "__main__":
    sys.exit(worker_main())
