from .client import JobException, JobInfo, JobState, SchedulerClient, make  # noqa: F401
