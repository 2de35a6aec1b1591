function matches = featureMatchingGlobal(input, allDescriptors, numImg)
    %FEATUREMATCHINGGLOBAL Shadows PP/featureMatching/featureMatchingGlobal.m (the default matcher, inputs.m:46).
    %   Float descriptors (SIFT, SURF): pooling, row normalisation, the exact k nearest of every pooled row and the
    %   per-query filter (drop self / same image, need two, ratio on squared L2; :69-161) run in one device pass through
    %   aps_mex('match_global') - an int8 proof pass dismisses the queries the filter provably drops, the others get
    %   their exact neighbours; the lists equal flann_knn_win + the reference's loop on an exact search.
    %   binaryFeatures descriptors and k > 4 are forwarded to the reference's own file (which then calls the
    %   flann_knn_win shadow for the Hamming k-NN).
    arguments
        input struct
        allDescriptors cell
        numImg (1, 1) {mustBeNumeric, mustBeFinite, mustBePositive}
    end
    if isempty(allDescriptors) || all(cellfun(@isempty, allDescriptors))
        matches = cell(numImg);
        return;
    end
    firstNonEmpty = find(~cellfun(@isempty, allDescriptors), 1, 'first');
    if isa(allDescriptors{firstNonEmpty}, 'binaryFeatures') || input.k > 4 || size(allDescriptors{firstNonEmpty}, 2) ~= 128
        matches = aps_call_shadowed('featureMatchingGlobal', mfilename('fullpath'), input, allDescriptors, numImg);
        return;
    end
    d = cellfun(@(x) single(x), allDescriptors(1:numImg), 'UniformOutput', false);
    matches = aps_mex('match_global', d, double(input.Ratiothreshold), double(input.k));
end
