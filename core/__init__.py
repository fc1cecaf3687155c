"""Reference-compatible import paths (`core.env`, `core.model`, `core.memory`,
`core.envs.*`, `core.models.*`, `core.memories.*`, `core.single_processes.*`):
thin aliases of the pytorch_distributed_amd package."""
