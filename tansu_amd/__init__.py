"""MI355X-native Kafka RecordBatch v2 codec (the record path of nisshi-sans-io)."""
