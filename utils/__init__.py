"""Reference-compatible import paths (`from utils.options import Options`,
`from utils.factory import ...`): thin aliases of pytorch_distributed_amd.utils."""
