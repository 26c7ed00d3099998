"""MI355X-native forward pass for llama2.ts."""
