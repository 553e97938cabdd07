from typing import Any

TaskType = Any


class TaskSettableEnv:
    def sample_tasks(self, n_tasks):
        raise NotImplementedError

    def set_task(self, task):
        raise NotImplementedError

    def get_task(self):
        raise NotImplementedError
